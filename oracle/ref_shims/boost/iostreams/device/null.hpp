#pragma once
#include <boost/iostreams/filtering_stream.hpp>
