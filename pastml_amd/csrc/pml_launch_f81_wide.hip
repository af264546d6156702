// Launcher of the F81-family level kernels for more than 256 states (round 6): one wavefront per unit, 64 lanes x 8 states
// (k <= 512), masks of five to eight words, 16-bit arg-max tables.  The reference has no bound on the number of states
// (pastml/ml.py:134); the other lane shapes stop at 256.  These contexts run the plain level schedule (pml_host.h, wide_states).
#include "pml_launch_f81_level.h"

int dispatch_sweep_f81_wide(pml_ctx* ctx, SweepKind what, const int* level, int n_level) {
    if (n_level <= 0) return PML_OK;
    launch_sweep_f81<64, 8>(ctx, what, level, n_level);
    HIP_TRY(hipGetLastError());
    return PML_OK;
}
