#include "petscsys_mock.h"
