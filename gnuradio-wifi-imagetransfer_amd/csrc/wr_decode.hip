#include "wr_kernels.h"
