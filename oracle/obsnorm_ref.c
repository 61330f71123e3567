/* TEST INFRASTRUCTURE — CPU restatement of the reference's running observation normaliser, the checker for
 * v4l_obs_norm. Never linked or loaded by the product.
 *
 * Follows torchrl/env/base_wrapper.py:
 *   :44-61  update_mean_var_count (the Welford/Chan merge "imported from OpenAI baselines")
 *   :64-72  Normalizer.__init__ (mean 0, var 1, count 1e-4, clip 10)
 *   :77-84  Normalizer.update_estimate: merge with np.mean(data, axis=0), np.var(data, axis=0), data.shape[0]
 *   :93-96  Normalizer.filt: clip((raw - mean) / (sqrt(var) + 1e-4), -clip, clip)
 * as NormObsWithImg.observation (vision4leg/get_env.py:58-67) and NormObs.observation (base_wrapper.py:119-122)
 * apply them to the [E][S] proprio block a vectorised env returns each step: update first (training mode only), then
 * filter with the updated statistics. numpy reduces axis 0 of a C-contiguous [E][S] array row after row (first row
 * copied, the others added in order), so every sum below runs e = 0..E-1 sequentially; np.var subtracts the
 * already-divided mean, squares, sums the same way and divides by E. Build with -ffp-contract=off (oracle/Makefile):
 * the bit pattern of every double is part of the contract. Pinned against the reference class itself by
 * tests/golden/make_golden_obsnorm.py -> tests/golden/obsnorm.npz. */
#include <math.h>

void obsnorm_ref(const double* raw, long ld_raw, int E, int S, double* mean, double* var, double* count, double clip,
                 int update, double* out, long ld_out) {
  const double cnt = *count, bc = (double)E;
  for (int d = 0; d < S; ++d) {
    double m = mean[d], v = var[d];
    if (update) {
      double s = raw[d];
      for (int e = 1; e < E; ++e) s = s + raw[(long)e * ld_raw + d];
      const double bm = s / bc; /* np.mean(data, axis=0) */
      double x = raw[d] - bm;
      double q = x * x;
      for (int e = 1; e < E; ++e) {
        x = raw[(long)e * ld_raw + d] - bm;
        q = q + x * x;
      }
      const double bv = q / bc; /* np.var(data, axis=0) */
      const double delta = bm - m;
      const double tot = cnt + bc;
      const double new_mean = m + delta * bc / tot;
      const double m_a = v * cnt;
      const double m_b = bv * bc;
      const double M2 = m_a + m_b + delta * delta * cnt * bc / tot;
      m = new_mean;
      v = M2 / tot;
      mean[d] = m;
      var[d] = v;
    }
    const double den = sqrt(v) + 1e-4;
    for (int e = 0; e < E; ++e) {
      double y = (raw[(long)e * ld_raw + d] - m) / den;
      y = y < -clip ? -clip : y; /* np.clip == minimum(maximum(y, lo), hi); NaN passes through both */
      y = y > clip ? clip : y;
      out[(long)e * ld_out + d] = y;
    }
  }
  if (update) *count = cnt + bc;
}
