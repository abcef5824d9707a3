// TEST INFRASTRUCTURE ONLY — stand-in for <opencv2/imgproc/imgproc_c.h>: the C-API constants vio.cpp names
#pragma once
#include <opencv2/opencv.hpp>
#define CV_GRAY2BGR 8
#define CV_BGR2GRAY 6
#define CV_INTER_LINEAR 1
#define CV_AA 16
