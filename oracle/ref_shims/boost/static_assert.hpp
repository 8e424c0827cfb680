#pragma once
#define BOOST_STATIC_ASSERT(x) static_assert(x, #x)
