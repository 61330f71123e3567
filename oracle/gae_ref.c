/* TEST INFRASTRUCTURE — plain-C restatement of the reference's GAE recursion. NOT part of the product
 * (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it).
 *
 * Follows torchrl/replay_buffers/on_policy.py:17-45 of /root/reference: float64 arithmetic, evaluated in the
 * numpy expression order, one rounding per operation (compile with -ffp-contract=off so gcc never fuses a
 * multiply-add — numpy does not). Arrays are [T][E] row-major (the reference's trailing axis of 1 dropped);
 * time_limits is [T] when tl_per_env == 0 (collector/on_policy.py:122-124 stores `[False]` rows) else [T][E].
 * Pinned against the reference itself by tests/golden/make_golden.py (bit-identical).
 */
#include <stddef.h>

void gae_ref(const double* rewards, const double* values, const double* terminals, const double* time_limits,
             int tl_per_env, const double* last_value, int T, int E, double gamma, double tau, int use_time_limit,
             double* advs, double* rets) {
  for (int e = 0; e < E; ++e) {
    double A = 0.0;
    double vnext = last_value[e];
    for (int t = T - 1; t >= 0; --t) {
      const size_t o = (size_t)t * E + e;
      const double c = (1.0 - terminals[o]) * gamma;   /* (1 - term) * gamma            */
      double delta = rewards[o] + c * vnext;           /* r + ((1-term)*gamma) * V[t+1] */
      delta = delta - values[o];                       /* ... - V[t]                    */
      A = delta + (c * tau) * A;                       /* delta + (((1-term)*gamma)*tau) * A */
      if (use_time_limit) A = A * (1.0 - time_limits[tl_per_env ? o : (size_t)t]);
      advs[o] = A;
      rets[o] = A + values[o];
      vnext = values[o];
    }
  }
}

/* discount_reward, torchrl/replay_buffers/on_policy.py:47-71 (PPO(gae=False)): same conventions as gae_ref.
 *   filter:  R = (r[t] + (1 - term[t]) * gamma * R * (1 - tl[t])) + tl[t] * V[t]      (python precedence: left to right)
 *   else:    R = r[t] + (1 - term[t]) * gamma * R
 *   advs[t] = R - V[t], rets[t] = R */
void discount_ref(const double* rewards, const double* values, const double* terminals, const double* time_limits,
                  int tl_per_env, const double* last_value, int T, int E, double gamma, int use_time_limit,
                  double* advs, double* rets) {
  for (int e = 0; e < E; ++e) {
    double R = last_value[e];
    for (int t = T - 1; t >= 0; --t) {
      const size_t o = (size_t)t * E + e;
      double x = (1.0 - terminals[o]) * gamma;
      x = x * R;
      if (use_time_limit) {
        const double tl = time_limits[tl_per_env ? o : (size_t)t];
        x = x * (1.0 - tl);
        x = rewards[o] + x;
        R = x + tl * values[o];
      } else {
        R = rewards[o] + x;
      }
      advs[o] = R - values[o];
      rets[o] = R;
    }
  }
}
