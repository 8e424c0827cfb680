#pragma once
#include <openbabel/mol.h>
