// TEST INFRASTRUCTURE ONLY — stand-in for <vikit/robust_cost.h>: included by the reference, nothing from it is used by the compiled sources
#pragma once
#include <Eigen/Dense>
