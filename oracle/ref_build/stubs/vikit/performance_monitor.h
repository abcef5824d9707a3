// TEST INFRASTRUCTURE ONLY — stand-in for <vikit/performance_monitor.h>: included by the reference, nothing from it is used by the compiled sources
#pragma once
#include <Eigen/Dense>
