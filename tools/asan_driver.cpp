// ASAN/UBSAN driver for the host-side analysis (order.cpp, symbolic.cpp, gen.cpp): random and time-coupled blocks
#include <algorithm>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <random>
#include <vector>
#include "common.h"
#include "pips_hip.h"
using namespace pips;
int main() {
   std::mt19937 rng(7);
   int n_done = 0;
   for (int rep = 0; rep < 60; ++rep) {
      const int nx = 50 + rng() % 3000, my = std::max(1, (int)(nx * (0.2 + 0.6 * (rng() % 100) / 100.0))), n0 = 1 + rng() % 12, myl = 1 + rng() % 12;
      const bool banded = rep % 2;
      const double rho = 6.0 / nx;
      const int kw = pips_gen_row_nnz(nx, rho);
      std::vector<int> Wrp(my + 1), Wci((size_t)my * kw), Trp(my + 1), Tci((size_t)my * 2), Frp(myl + 1), Fci((size_t)myl * 4 + 4);
      std::vector<double> Wv((size_t)my * kw), Tv((size_t)my * 2), Fv((size_t)myl * 4 + 4), c(nx), xs(nx);
      if (pips_gen_block(11, rep + 1, nx, my, std::max(n0, 1), std::max(myl, 1), rho, Wrp.data(), Wci.data(), Wv.data(), Trp.data(), Tci.data(), Tv.data(),
                         Frp.data(), Fci.data(), Fv.data(), c.data(), xs.data())) { printf("gen failed: %s\n", pips_hip_last_error()); return 1; }
      if (banded) {   // overwrite W with a band
         int p = 0;
         for (int r = 0; r < my; ++r) {
            Wrp[r] = p;
            const int center = (int)((long long)r * nx / my);
            std::vector<int> cols;
            for (int k = -2; k <= 2; ++k) { int cc = center + k * (1 + (int)(rng() % 5)); if (cc >= 0 && cc < nx) cols.push_back(cc); }
            std::sort(cols.begin(), cols.end()); cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
            for (int cc : cols) { if (p < (int)Wci.size()) { Wci[p] = cc; Wv[p] = 1.0; ++p; } }
         }
         Wrp[my] = p;
      }
      const int n = nx + my;
      std::vector<int> Krp(n + 1), dpos(n);
      pips_kkt_leaf_assemble(nx, my, 0, nullptr, nullptr, nullptr, Wrp.data(), Wci.data(), Wv.data(), nullptr, nullptr, nullptr, Krp.data(), nullptr, nullptr, nullptr);
      std::vector<int> Kci(Krp[n]); std::vector<double> Kv(Krp[n]);
      pips_kkt_leaf_assemble(nx, my, 0, nullptr, nullptr, nullptr, Wrp.data(), Wci.data(), Wv.data(), nullptr, nullptr, nullptr, Krp.data(), Kci.data(), Kv.data(), dpos.data());
      CsrPattern K{n, n, Krp.data(), Kci.data()};
      CsrPattern B{0, n, nullptr, nullptr};
      for (int variant = 0; variant < 3; ++variant) {
         AnalyzeOptions opt;
         opt.nd_depth = variant == 0 ? 0 : 4;
         opt.relax_zeros = variant == 2 ? 0.0 : 0.4;
         opt.force_n_head = variant == 1 ? n : -1;
         BlockSym sym;
         if (analyze_block(K, B, nx, opt, sym)) { printf("analyze failed: %s\n", last_error()); return 1; }
         long long cc = 0; for (int v : sym.colcount) cc += v;
         if ((int)sym.perm.size() != n) { printf("bad perm\n"); return 1; }
         ++n_done;
      }
   }
   printf("analysed %d block variants under ASAN/UBSAN\n", n_done);
   return 0;
}
namespace pips {
static thread_local std::string g_err;
void set_last_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
}
extern "C" const char* pips_hip_last_error(void) { return pips::last_error(); }
