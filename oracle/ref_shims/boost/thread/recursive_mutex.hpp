#pragma once
#include <boost/thread/mutex.hpp>
