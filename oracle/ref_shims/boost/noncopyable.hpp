#pragma once
#include <boost/utility.hpp>
