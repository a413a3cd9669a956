// Sanitizer driver for the GDX reader (csrc/gdx.cpp): reads every file given on the command line as each of its blocks, then
// hammers the parser with damaged copies - truncations of the first file (every length at first, sparser further in) and 500 random
// byte flips / overwrites of all files.  A damaged file may be rejected or may parse to different numbers; it must never read
// or write out of bounds (AddressSanitizer / UBSan abort the process if it does).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "pips_hip.h"

// the error slot lives in engine.hip (device code) in the library; this host-only build supplies it
static thread_local std::string g_err;
namespace pips {
void set_last_error(const std::string& m) { g_err = m; }
const char* last_error() { return g_err.c_str(); }
}
extern "C" const char* pips_hip_last_error(void) { return g_err.c_str(); }

static std::vector<unsigned char> slurp(const char* path) {
   std::vector<unsigned char> d;
   FILE* f = fopen(path, "rb");
   if (!f) return d;
   fseek(f, 0, SEEK_END);
   long n = ftell(f);
   fseek(f, 0, SEEK_SET);
   d.resize(n > 0 ? n : 0);
   if (!d.empty() && fread(d.data(), 1, d.size(), f) != d.size()) d.clear();
   fclose(f);
   return d;
}

static void dump(const char* path, const std::vector<unsigned char>& d, size_t n) {
   FILE* f = fopen(path, "wb");
   if (!f) exit(3);
   if (n) fwrite(d.data(), 1, n, f);
   fclose(f);
}

static int read_all(const char* path, int nblocks, long long* checksum) {
   int ok = 0;
   for (int k = 0; k < nblocks; ++k) {
      void* b = nullptr;
      if (pips_gdx_read_block(&b, path, nblocks, k, 1) != 0) continue;
      ++ok;
      long long cnt[14];
      pips_gdx_block_counts(b, cnt);
      for (int w = 0; w < 15; ++w) {
         int len = 0;
         pips_gdx_block_vector(b, w, nullptr, 0, &len);
         std::vector<double> v(len + 1);
         pips_gdx_block_vector(b, w, v.data(), len, &len);
         *checksum += len;
      }
      for (int w = 0; w < 6; ++w) {
         int present = 0, rows = 0, cols = 0;
         pips_gdx_block_matrix(b, w, &present, &rows, &cols, nullptr, nullptr, nullptr);
         std::vector<int> rp(rows + 1), ci(cnt[6 + w] + 1);
         std::vector<double> va(cnt[6 + w] + 1);
         pips_gdx_block_matrix(b, w, nullptr, nullptr, nullptr, rp.data(), ci.data(), va.data());
         *checksum += rp[rows];
      }
      pips_gdx_block_destroy(b);
   }
   return ok;
}

int main(int argc, char** argv) {
   if (argc < 4) { printf("usage: asan_gdx_driver <scratch file> <nblocks> <gdx file>...\n"); return 2; }
   const char* scratch = argv[1];
   const int nblocks = atoi(argv[2]);
   std::mt19937 rng(99);
   long long checksum = 0;
   int n_valid = 0, n_damaged = 0, n_damaged_ok = 0;
   for (int a = 3; a < argc; ++a) {
      std::vector<unsigned char> d = slurp(argv[a]);
      if (d.empty()) { printf("cannot read %s\n", argv[a]); return 1; }
      if (read_all(argv[a], nblocks, &checksum) != nblocks) { printf("valid file %s rejected: %s\n", argv[a], pips_hip_last_error()); return 1; }
      ++n_valid;
      if (a == 3)
         for (size_t n = 0; n < d.size() && n < 4096; n += 1 + n / 256) {
            dump(scratch, d, n);
            n_damaged_ok += read_all(scratch, nblocks, &checksum) > 0;
            ++n_damaged;
         }
      for (int rep = 0; rep < 500; ++rep) {
         std::vector<unsigned char> e = d;
         const int hits = 1 + rng() % 4;
         for (int h = 0; h < hits; ++h) {
            const size_t at = rng() % e.size();
            if (rng() % 2) e[at] ^= (unsigned char)(1u << (rng() % 8));
            else e[at] = (unsigned char)(rng() % 256);
         }
         dump(scratch, e, e.size());
         n_damaged_ok += read_all(scratch, nblocks, &checksum) > 0;
         ++n_damaged;
      }
   }
   printf("gdx reader: %d valid files, %d damaged variants (%d still parsed), checksum %lld\n", n_valid, n_damaged, n_damaged_ok, checksum);
   return 0;
}
