// ssm/common_headers.h -- counterpart of the reference's include/common_headers.h (std + third-party includes, ANSI colours)
#pragma once
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include "compat.h"
using namespace std;      // the reference's headers do this (include/common_headers.h); kept so its driver bodies compile
#define RESET "\033[0m"
#define RED "\033[31m"
#define GREEN "\033[32m"
#define YELLOW "\033[33m"
