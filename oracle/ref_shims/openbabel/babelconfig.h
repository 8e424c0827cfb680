#pragma once
#include <openbabel/mol.h>
#define OB_VERSION 0x030100
#define OB_VERSION_CHECK(a, b, c) (((a) << 16) + ((b) << 8) + (c))
