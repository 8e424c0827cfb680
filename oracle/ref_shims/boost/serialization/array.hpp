#pragma once
#include <boost/array.hpp>
