#pragma once
#include <boost/unordered_map.hpp>
