/* kdtree.c — TEST INFRASTRUCTURE (oracle); see sf_oracle.h and kdtree_impl.inc. */
#include "sf_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>

#define REAL float
#define REAL_MAX FLT_MAX
#define SUF f
#include "kdtree_impl.inc"
#undef REAL
#undef REAL_MAX
#undef SUF

#define REAL double
#define REAL_MAX DBL_MAX
#define SUF d
#include "kdtree_impl.inc"
#undef REAL
#undef REAL_MAX
#undef SUF

int orc_kdtree_f_size(const orc_kdtree_f *t) { return t ? t->n : 0; }

void orc_bruteforce_nn_f(const float *tgt, int n, const float *q, int m, int *idx, float *d2)
{
    for (int i = 0; i < m; ++i) {
        const float *a = q + 3 * (size_t)i;
        float worst = FLT_MAX;
        int best = -1;
        if (isfinite(a[0]) && isfinite(a[1]) && isfinite(a[2])) {
            for (int j = 0; j < n; ++j) {
                const float *p = tgt + 3 * (size_t)j;
                if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
                float acc = 0.f, diff;
                diff = a[0] - p[0]; acc += diff * diff;
                diff = a[1] - p[1]; acc += diff * diff;
                diff = a[2] - p[2]; acc += diff * diff;
                if (acc < worst) { worst = acc; best = j; }
            }
        }
        idx[i] = best;
        d2[i] = best >= 0 ? worst : INFINITY;
    }
}
