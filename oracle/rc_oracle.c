/* CPU ORACLE (test infrastructure only; see rc_oracle_impl.h for the
 * reference file:line each routine follows).  Built by oracle/Makefile into
 * oracle/_build/librc_oracle.so and loaded through ctypes by oracle/c_oracle.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define T double
#define SUF d
#define FABS fabs
#define SQRT sqrt
#define HYPOT hypot
#define COPYSIGN copysign
#define EPS_HALF 1.1102230246251565e-16 /* dlamch('Epsilon') = 2^-53 */
#include "rc_oracle_impl.h"
#undef T
#undef SUF
#undef FABS
#undef SQRT
#undef HYPOT
#undef COPYSIGN
#undef EPS_HALF

#define T float
#define SUF s
#define FABS fabsf
#define SQRT sqrtf
#define HYPOT hypotf
#define COPYSIGN copysignf
#define EPS_HALF 5.9604644775390625e-08f /* slamch('Epsilon') = 2^-24 */
#include "rc_oracle_impl.h"
