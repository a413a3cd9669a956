#include <stdio.h>
#include <stdlib.h>
#include "gmspipsio.h"
/* Test driver for read_block_compat.c: reads block argv[3] of argv[2] from the GDX file argv[1] through readBlock (reference signature)
 * and prints every field of GMSPIPSBlockData_t, one "name: values" line each. */
static void pd(const char* n, const double* v, long long len) { printf("%s:", n); for (long long i = 0; v && i < len; ++i) printf(" %.17g", v[i]); printf("\n"); }
static void pi16(const char* n, const int16_t* v, long long len) { printf("%s:", n); for (long long i = 0; v && i < len; ++i) printf(" %d", (int)v[i]); printf("\n"); }
static void pi32(const char* n, const int32_t* v, long long len) { printf("%s:", n); for (long long i = 0; v && i < len; ++i) printf(" %d", (int)v[i]); printf("\n"); }
int main(int argc, char** argv) {
   GMSPIPSBlockData_t blk;
   if (readBlock(atoi(argv[2]), atoi(argv[3]), 0, 1, argv[1], NULL, &blk)) return 1;
   printf("counts: %d %d %d %d %d %d %lld %lld %lld %lld %lld %lld\n", blk.n0, blk.ni, blk.mA, blk.mC, blk.mBL, blk.mDL, (long long)blk.nnzA, (long long)blk.nnzB, (long long)blk.nnzC, (long long)blk.nnzD, (long long)blk.nnzBL, (long long)blk.nnzDL);
   pd("c", blk.c, blk.ni); pd("xlow", blk.xlow, blk.ni); pd("xupp", blk.xupp, blk.ni); pi16("ixlow", blk.ixlow, blk.ni); pi16("ixupp", blk.ixupp, blk.ni);
   pd("b", blk.b, blk.mA); pd("clow", blk.clow, blk.mC); pd("cupp", blk.cupp, blk.mC); pi16("iclow", blk.iclow, blk.mC); pi16("icupp", blk.icupp, blk.mC);
   pd("bL", blk.bL, blk.mBL); pd("dlow", blk.dlow, blk.mDL); pd("dupp", blk.dupp, blk.mDL); pi16("idlow", blk.idlow, blk.mDL); pi16("idupp", blk.idupp, blk.mDL);
   pi32("rmA", blk.rmA, blk.rmA ? blk.mA + 1 : 0); pi32("ciA", blk.ciA, blk.rmA ? blk.nnzA : 0); pd("valA", blk.valA, blk.rmA ? blk.nnzA : 0);
   pi32("rmB", blk.rmB, blk.rmB ? blk.mA + 1 : 0); pi32("ciB", blk.ciB, blk.rmB ? blk.nnzB : 0); pd("valB", blk.valB, blk.rmB ? blk.nnzB : 0);
   pi32("rmC", blk.rmC, blk.rmC ? blk.mC + 1 : 0); pi32("ciC", blk.ciC, blk.rmC ? blk.nnzC : 0); pd("valC", blk.valC, blk.rmC ? blk.nnzC : 0);
   pi32("rmD", blk.rmD, blk.rmD ? blk.mC + 1 : 0); pi32("ciD", blk.ciD, blk.rmD ? blk.nnzD : 0); pd("valD", blk.valD, blk.rmD ? blk.nnzD : 0);
   pi32("rmBL", blk.rmBL, blk.rmBL ? blk.mBL + 1 : 0); pi32("ciBL", blk.ciBL, blk.rmBL ? blk.nnzBL : 0); pd("valBL", blk.valBL, blk.rmBL ? blk.nnzBL : 0);
   pi32("rmDL", blk.rmDL, blk.rmDL ? blk.mDL + 1 : 0); pi32("ciDL", blk.ciDL, blk.rmDL ? blk.nnzDL : 0); pd("valDL", blk.valDL, blk.rmDL ? blk.nnzDL : 0);
   return 0;
}
