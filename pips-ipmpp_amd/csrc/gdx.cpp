// Input files of the reference: one "jacobian" GDX file per block (gmspips_reader, Drivers/gams/gmspips/gmspips_reader.cpp:
// 30-60 -> readBlock, gmspipsio.c:1357-2033, which goes through the GAMS GDX library).  This file restates the two layers
// that are needed and nothing else of that library:
//   GdxFile     the container (version 7, uncompressed): stream signature, header with the section offsets, symbol table,
//               per-symbol record stream - a first-changed-dimension byte, delta-coded keys whose width follows the key
//               range of the dimension, values coded as one byte (0 undefined, 1 NA, 2 +inf, 3 -inf, 4 EPS, 5 zero, 6 one,
//               7 minus one, 8 one half, 9 two, 10 "a double follows");
//   read_block  the block extraction rules of readBlock: variables / equations of stage (= scale field) k - offset belong
//               to block k, equations of stage num_blocks + offset are linking rows, both bounds finite = equality,
//               one infinite = inequality, free rows and the objective row are dropped, c = -direction * coef / objcoef.
// Host code only (no device work); pips-ipmpp_amd/python/gdx.py is the same logic in Python and the tests compare the two.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <strings.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"
#include "pips_hip.h"

namespace pips {
namespace {

constexpr double SV_UNDEF = 1.0e300, SV_NA = 2.0e300, SV_PINF = 3.0e300, SV_MINF = 4.0e300;
constexpr int MARK_BOI = 19510624;
enum { GDX_SET = 0, GDX_PARAMETER = 1, GDX_VARIABLE = 2, GDX_EQUATION = 3 };
enum { V_LEVEL = 0, V_MARGINAL = 1, V_LOWER = 2, V_UPPER = 3, V_SCALE = 4 };

struct Cursor {
   const std::vector<unsigned char>& d;
   size_t p;
   bool ok = true;
   Cursor(const std::vector<unsigned char>& data, size_t pos) : d(data), p(pos) {}
   template <class T>
   T take() {
      T v{};
      if (p + sizeof(T) > d.size()) { ok = false; return v; }
      std::memcpy(&v, d.data() + p, sizeof(T));
      p += sizeof(T);
      return v;
   }
   std::string str() {
      const unsigned n = take<uint8_t>();
      if (!ok || p + n > d.size()) { ok = false; return std::string(); }
      std::string s((const char*)d.data() + p, n);
      p += n;
      return s;
   }
};

struct Symbol {
   std::string name;
   int64_t pos = 0;
   int dim = 0, type = 0, count = 0;
   std::vector<int> keys;       // count x dim
   std::vector<double> vals;    // count x nvals
   int nvals = 1;
   bool loaded = false;
};

struct GdxFile {
   std::vector<unsigned char> data;
   std::vector<Symbol> syms;
   std::string err;

   bool fail(const std::string& m) { err = m; return false; }

   bool open(const char* path) {
      FILE* f = std::fopen(path, "rb");
      if (!f) return fail(std::string("cannot open ") + path);
      std::fseek(f, 0, SEEK_END);
      const long n = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      data.resize(n > 0 ? (size_t)n : 0);
      const size_t got = data.empty() ? 0 : std::fread(data.data(), 1, data.size(), f);
      std::fclose(f);
      if (got != data.size()) return fail("short read");
      Cursor c(data, 0);
      if (c.take<uint8_t>() != 2 || c.take<uint16_t>() != 0x1234 || c.take<uint8_t>() != 4 || c.take<uint32_t>() != 0x12345678u ||
          c.take<uint8_t>() != 8 || !c.ok)
         return fail("not a little-endian GDX stream");
      c.take<double>();
      if (c.take<uint8_t>() != 123 || c.str() != "GAMSGDX" || !c.ok) return fail("GDX header not found");
      const int version = c.take<int32_t>(), compressed = c.take<int32_t>();
      if (version != 7 || compressed != 0) return fail("only uncompressed GDX version 7 files are supported");
      c.str();   // audit line
      c.str();   // producer
      if (c.take<int32_t>() != MARK_BOI || !c.ok) return fail("index marker not found");
      const int64_t sym_pos = c.take<int64_t>();
      if (!c.ok || sym_pos <= 0 || (size_t)sym_pos >= data.size()) return fail("bad symbol table offset");
      Cursor s(data, (size_t)sym_pos);
      if (s.str() != "_SYMB_") return fail("symbol table marker not found");
      const int nsym = s.take<int32_t>();
      for (int i = 0; i < nsym && s.ok; ++i) {
         Symbol y;
         y.name = s.str();
         y.pos = s.take<int64_t>();
         y.dim = s.take<int32_t>();
         y.type = s.take<uint8_t>();
         s.take<int32_t>();   // user info
         y.count = s.take<int32_t>();
         s.take<int32_t>();   // error count
         s.take<uint8_t>();   // has set text
         s.str();             // explanatory text
         if (s.take<uint8_t>() != 0) return fail("compressed records are not supported");
         if (s.take<uint8_t>() != 0)
            for (int d = 0; d < y.dim; ++d) s.take<int32_t>();
         const int ncomment = s.take<int32_t>();
         for (int k = 0; k < ncomment && s.ok; ++k) s.str();
         y.nvals = (y.type == GDX_VARIABLE || y.type == GDX_EQUATION) ? 5 : 1;
         syms.push_back(std::move(y));
      }
      if (!s.ok) return fail("truncated symbol table");
      return true;
   }

   Symbol* find(const char* name) {
      for (auto& y : syms)
         if (strcasecmp(y.name.c_str(), name) == 0) return load(y) ? &y : nullptr;
      err = std::string("symbol ") + name + " not in file";
      return nullptr;
   }

   bool load(Symbol& y) {
      if (y.loaded) return true;
      if (y.type > GDX_EQUATION || y.dim < 0 || y.dim > 20) return fail("symbol " + y.name + ": unsupported type or dimension");
      if (y.pos <= 0 || (size_t)y.pos >= data.size()) return fail("symbol " + y.name + ": bad data offset");
      Cursor c(data, (size_t)y.pos);
      if (c.str() != "_DATA_") return fail("symbol " + y.name + ": data marker not found");
      if (c.take<uint8_t>() != y.dim) return fail("symbol " + y.name + ": dimension mismatch");
      c.take<int32_t>();
      std::vector<int> lo(y.dim), width(y.dim);
      for (int d = 0; d < y.dim; ++d) {
         lo[d] = c.take<int32_t>();
         const int64_t span = (int64_t)c.take<int32_t>() - lo[d];
         width[d] = span <= 255 ? 1 : (span <= 65535 ? 2 : 4);
      }
      std::vector<int> key(y.dim, 0);
      static const double coded[10] = {SV_UNDEF, SV_NA, SV_PINF, SV_MINF, 0.0 /* EPS reads as 0, readBlock :1420-1425 */, 0.0, 1.0, -1.0, 0.5, 2.0};
      while (c.ok) {
         const int b = c.take<uint8_t>();
         if (b == 255) break;
         if (y.dim > 0) {
            constexpr int64_t KEY_MAX = 2000000000;   // label numbers are positive 32-bit integers
            if (b > y.dim) {
               const int64_t kv = (int64_t)key[y.dim - 1] + (b - y.dim);
               if (kv > KEY_MAX) return fail("symbol " + y.name + ": key out of range");
               key[y.dim - 1] = (int)kv;
            }
            else if (b >= 1)
               for (int d = b - 1; d < y.dim; ++d) {
                  const int64_t kv = (int64_t)lo[d] + (width[d] == 1 ? (int64_t)c.take<uint8_t>() : (width[d] == 2 ? (int64_t)c.take<uint16_t>() : (int64_t)c.take<int32_t>()));
                  if (kv < -KEY_MAX || kv > KEY_MAX) return fail("symbol " + y.name + ": key out of range");
                  key[d] = (int)kv;
               }
            else return fail("symbol " + y.name + ": bad record header");
         }
         for (int v = 0; v < y.nvals; ++v) {
            const int code = c.take<uint8_t>();
            if (code > 10) return fail("symbol " + y.name + ": bad value code");
            y.vals.push_back(code == 10 ? c.take<double>() : coded[code]);
         }
         y.keys.insert(y.keys.end(), key.begin(), key.end());
      }
      if (!c.ok) return fail("symbol " + y.name + ": truncated records");
      const int n = (int)(y.vals.size() / y.nvals);
      if (y.count >= 0 && y.count != n) return fail("symbol " + y.name + ": record count differs from the symbol table");
      y.count = n;
      y.loaded = true;
      return true;
   }
};

// stage / direction fields are doubles in the file: anything that is not a small integer maps to a value that matches no block
inline int to_int(double v) { return (v > -1.0e9 && v < 1.0e9) ? (int)v : -1000000000; }

struct CsrRows {
   int rows = 0, cols = 0;
   bool present = false;
   std::vector<std::vector<std::pair<int, double>>> r;
};

struct Block {
   long long counts[14] = {0};
   std::vector<double> vec[15];
   CsrRows mat[6];   // A B C D BL DL
};

}  // namespace

static int read_block(const char* path, int num_blocks, int act_block, int offset, Block& out, std::string& err) {
   GdxFile g;
   if (!g.open(path)) { err = g.err; return PIPS_ERR_ARG; }
   Symbol *objcoef = g.find("objcoef"), *jobj = g.find("jobj"), *jset = g.find("j"), *iset = g.find("i"), *x = g.find("x"), *e = g.find("e"),
          *A = g.find("A");
   if (!objcoef || !jobj || !jset || !iset || !x || !e || !A) { err = g.err; return PIPS_ERR_ARG; }
   if (objcoef->count < 1 || jobj->count < 1 || jobj->dim != 1 || jset->dim != 1 || iset->dim != 1 || x->dim != 1 || e->dim != 1 || A->dim != 2) {
      err = "not a jacobian GDX file (symbol shapes)";
      return PIPS_ERR_ARG;
   }
   const int direction = to_int(objcoef->vals[V_LEVEL]);
   if (direction != 1 && direction != -1) { err = "objcoef must be 1 (min) or -1 (max)"; return PIPS_ERR_ARG; }
   const int obj_var = jobj->keys[0];
   std::map<int, int> col_of, row_of;
   for (int n = 0; n < jset->count; ++n) col_of[jset->keys[n]] = n;
   for (int m = 0; m < iset->count; ++m) row_of[iset->keys[m]] = m;
   const int gdx_n = jset->count, gdx_m = iset->count;
   auto col = [&](int label, int* c) { auto it = col_of.find(label); if (it == col_of.end()) return false; *c = it->second; return true; };
   auto row = [&](int label, int* r) { auto it = row_of.find(label); if (it == row_of.end()) return false; *r = it->second; return true; };
   // ---- variables
   std::vector<int> var_perm(gdx_n, 0);
   int n0 = 0, ni = 0;
   for (int r = 0; r < x->count; ++r) {
      const int k = x->keys[r];
      if (k == obj_var) continue;
      int cj;
      if (!col(k, &cj)) { err = "variable record without a label in set j"; return PIPS_ERR_ARG; }
      const int blk = to_int(x->vals[5 * r + V_SCALE]) - offset;
      if (blk == 0) { var_perm[cj] = 1; ++n0; }
      else if (blk == act_block) { var_perm[cj] = 2; ++ni; }
   }
   for (int j = 0, c0 = 0, ci = 0; j < gdx_n; ++j) {
      if (var_perm[j] == 1) var_perm[j] = ++c0;
      else if (var_perm[j] == 2) var_perm[j] = n0 + ++ci;
   }
   if (act_block == 0) ni = n0;
   std::vector<double>&c = out.vec[0], &xlow = out.vec[1], &xupp = out.vec[2], &ixlow = out.vec[3], &ixupp = out.vec[4];
   c.assign(ni, 0.0); xlow.assign(ni, 0.0); xupp.assign(ni, 0.0); ixlow.assign(ni, 0.0); ixupp.assign(ni, 0.0);
   for (int r = 0, n = 0; r < x->count; ++r) {
      if (x->keys[r] == obj_var || to_int(x->vals[5 * r + V_SCALE]) - offset != act_block) continue;
      if (n >= ni) { err = "variable count mismatch"; return PIPS_ERR_ARG; }
      if (x->vals[5 * r + V_LOWER] != SV_MINF) { xlow[n] = x->vals[5 * r + V_LOWER]; ixlow[n] = 1.0; }
      if (x->vals[5 * r + V_UPPER] != SV_PINF) { xupp[n] = x->vals[5 * r + V_UPPER]; ixupp[n] = 1.0; }
      ++n;
   }
   // ---- objective row
   int obj_row = 0;
   double obj_coef = 0.0;
   for (int r = 0; r < A->count; ++r)
      if (A->keys[2 * r + 1] == obj_var) {
         if (obj_row) { err = "objective variable used in more than one row"; return PIPS_ERR_ARG; }
         obj_row = A->keys[2 * r];
         obj_coef = A->vals[r];
      }
   if (obj_row && obj_coef == 0.0) { err = "zero coefficient of the objective variable"; return PIPS_ERR_ARG; }
   for (int r = 0; r < A->count && obj_row; ++r) {
      if (A->keys[2 * r] != obj_row || A->keys[2 * r + 1] == obj_var) continue;
      int cj;
      if (!col(A->keys[2 * r + 1], &cj)) { err = "matrix column without a label in set j"; return PIPS_ERR_ARG; }
      const int p = var_perm[cj];
      if (p == 0 || (p <= n0 && act_block > 0)) continue;
      c[p - (p <= n0 ? 1 : n0 + 1)] = direction * (-A->vals[r] / obj_coef);
   }
   // ---- equations: 1 A, 2 C, 3 BL, 4 DL
   std::vector<int> etype(gdx_m, 0);
   std::vector<double>&b = out.vec[5], &clow = out.vec[6], &cupp = out.vec[7], &iclow = out.vec[8], &icupp = out.vec[9], &bL = out.vec[10],
                      &dlow = out.vec[11], &dupp = out.vec[12], &idlow = out.vec[13], &idupp = out.vec[14];
   for (int r = 0; r < e->count; ++r) {
      const int k = e->keys[r];
      const double lo = e->vals[5 * r + V_LOWER], up = e->vals[5 * r + V_UPPER];
      const bool lo_inf = lo == SV_MINF, up_inf = up == SV_PINF;
      if ((lo_inf && up_inf) || k == obj_row) continue;
      const int blk = to_int(e->vals[5 * r + V_SCALE]) - offset;
      if (blk != act_block && blk != num_blocks) continue;
      const bool link = blk == num_blocks, ineq = lo_inf || up_inf;
      int m;
      if (!row(k, &m)) { err = "equation record without a label in set i"; return PIPS_ERR_ARG; }
      etype[m] = link ? (ineq ? 4 : 3) : (ineq ? 2 : 1);
      std::vector<double>&lows = link ? dlow : clow, &upps = link ? dupp : cupp, &il = link ? idlow : iclow, &iu = link ? idupp : icupp;
      if (lo_inf) { lows.push_back(0.0); il.push_back(0.0); upps.push_back(up); iu.push_back(1.0); }
      else if (up_inf) { lows.push_back(lo); il.push_back(1.0); upps.push_back(0.0); iu.push_back(0.0); }
      else (link ? bL : b).push_back(lo);
   }
   int m_of[5] = {0, 0, 0, 0, 0};
   std::vector<int> pos_in_class(gdx_m, 0);
   for (int m = 0; m < gdx_m; ++m)
      if (etype[m]) pos_in_class[m] = m_of[etype[m]]++;
   const int nloc = act_block == 0 ? n0 : ni;
   const int mrows[6] = {m_of[1], m_of[1], m_of[2], m_of[2], m_of[3], m_of[4]};
   const int mcols[6] = {n0, ni, n0, ni, nloc, nloc};
   const bool present[6] = {m_of[1] > 0, m_of[1] > 0 && act_block != 0, m_of[2] > 0, m_of[2] > 0 && act_block != 0, m_of[3] > 0, m_of[4] > 0};
   for (int q = 0; q < 6; ++q) {
      out.mat[q].rows = mrows[q]; out.mat[q].cols = mcols[q]; out.mat[q].present = present[q];
      out.mat[q].r.assign(mrows[q], {});
   }
   for (int r = 0; r < A->count; ++r) {
      const int ri = A->keys[2 * r], cj = A->keys[2 * r + 1];
      if (ri == obj_row) continue;
      int rw, cl;
      if (!row(ri, &rw) || !col(cj, &cl)) { err = "matrix entry without labels in i / j"; return PIPS_ERR_ARG; }
      const int t = etype[rw], p = var_perm[cl];
      if (t == 0 && (p == 0 || p <= n0)) continue;
      if (t > 2 && (p == 0 || (p <= n0 && act_block != 0))) continue;
      if (t == 0 || p == 0) { err = "unexpected matrix coefficient: row and column belong to different blocks"; return PIPS_ERR_ARG; }
      const int which = p <= n0 ? (t == 1 ? 0 : (t == 2 ? 2 : (t == 3 ? 4 : 5))) : (t == 1 ? 1 : (t == 2 ? 3 : (t == 3 ? 4 : 5)));
      out.mat[which].r[pos_in_class[rw]].push_back({p <= n0 ? p - 1 : p - n0 - 1, A->vals[r]});
   }
   long long nnz[6];
   for (int q = 0; q < 6; ++q) {
      nnz[q] = 0;
      for (auto& rr : out.mat[q].r) nnz[q] += (long long)rr.size();
   }
   const long long cnt[14] = {n0, ni, m_of[1], m_of[2], m_of[3], m_of[4], nnz[0], nnz[1], nnz[2], nnz[3], nnz[4], nnz[5], num_blocks, act_block};
   std::memcpy(out.counts, cnt, sizeof(cnt));
   return PIPS_OK;
}

}  // namespace pips

using namespace pips;

extern "C" {

int pips_gdx_read_block(void** block, const char* path, int num_blocks, int act_block, int offset) {
   if (!block || !path || num_blocks <= 0 || num_blocks > 100000000 || act_block < 0 || act_block >= num_blocks || offset < -1000000 || offset > 1000000)
      PIPS_FAIL(PIPS_ERR_ARG, "pips_gdx_read_block: bad arguments");
   auto b = std::make_unique<Block>();
   std::string err;
   const int rc = read_block(path, num_blocks, act_block, offset, *b, err);
   if (rc) PIPS_FAIL(rc, "pips_gdx_read_block(%s): %s", path, err.c_str());
   *block = b.release();
   return PIPS_OK;
}

int pips_gdx_block_counts(void* block, long long* counts14) {
   if (!block || !counts14) PIPS_FAIL(PIPS_ERR_ARG, "pips_gdx_block_counts: bad arguments");
   std::memcpy(counts14, ((Block*)block)->counts, sizeof(((Block*)block)->counts));
   return PIPS_OK;
}

int pips_gdx_block_vector(void* block, int which, double* out, int capacity, int* length) {
   if (!block || which < 0 || which >= 15 || !length) PIPS_FAIL(PIPS_ERR_ARG, "pips_gdx_block_vector: bad arguments");
   const std::vector<double>& v = ((Block*)block)->vec[which];
   *length = (int)v.size();
   if (out) {
      if (capacity < (int)v.size()) PIPS_FAIL(PIPS_ERR_ARG, "pips_gdx_block_vector: buffer too small");
      if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(double));
   }
   return PIPS_OK;
}

int pips_gdx_block_matrix(void* block, int which, int* present, int* rows, int* cols, int* rowptr, int* colidx, double* val) {
   if (!block || which < 0 || which >= 6) PIPS_FAIL(PIPS_ERR_ARG, "pips_gdx_block_matrix: bad arguments");
   const CsrRows& m = ((Block*)block)->mat[which];
   if (present) *present = m.present ? 1 : 0;
   if (rows) *rows = m.rows;
   if (cols) *cols = m.cols;
   int p = 0;
   if (rowptr) rowptr[0] = 0;
   for (int r = 0; r < m.rows; ++r) {
      for (const auto& cv : m.r[r]) {
         if (colidx) colidx[p] = cv.first;
         if (val) val[p] = cv.second;
         ++p;
      }
      if (rowptr) rowptr[r + 1] = p;
   }
   return PIPS_OK;
}

void pips_gdx_block_destroy(void* block) { delete (Block*)block; }

}  // extern "C"
