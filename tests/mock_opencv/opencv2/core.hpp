// tests only: the minimal cv::Mat of opencv.hpp (see there)
#include "opencv.hpp"
