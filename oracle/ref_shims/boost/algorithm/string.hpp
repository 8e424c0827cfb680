#pragma once
#include <boost/algorithm/string/predicate.hpp>
