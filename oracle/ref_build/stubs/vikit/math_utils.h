// TEST INFRASTRUCTURE ONLY — stand-in for <vikit/math_utils.h>: included by the reference, nothing from it is used by the compiled sources
#pragma once
#include <Eigen/Dense>
