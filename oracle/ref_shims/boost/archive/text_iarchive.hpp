#pragma once
#include <boost/archive/inert.hpp>
