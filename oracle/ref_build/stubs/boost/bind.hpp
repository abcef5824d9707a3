// TEST INFRASTRUCTURE ONLY — stand-in for <boost/bind.hpp> (included by frame.cpp, unused)
#pragma once
