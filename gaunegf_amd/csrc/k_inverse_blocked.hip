// placeholder until the blocked MFMA Gauss-Jordan kernel lands
#include "negf_common.h"
bool inverse_blocked_supported(int) { return false; }
void launch_inverse_blocked(hipStream_t, int, int, cplx*, int*, int*) {}
