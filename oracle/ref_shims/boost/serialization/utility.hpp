#pragma once
#include <boost/serialization/access.hpp>
