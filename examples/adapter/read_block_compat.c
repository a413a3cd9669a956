/* Drop-in for the reference's readBlock (Drivers/gams/gmspips/gmspipsio.h:81-87, implementation gmspipsio.c:1357-2033, which
 * needs the GAMS GDX library): same signature, same GMSPIPSBlockData_t fields, filled through this library's GDX reader
 * (pips_gdx_read_block and accessors).  Compiled against the reference's own gmspipsio.h - nothing of the reference is copied
 * here; tests/test_adapter_compiles.py builds it where the reference tree is present, reads the reference's block files through
 * it and compares every field with the Python reader.  freeBlock of the reference (gmspipsio.c:400-486) releases what is
 * allocated here (plain malloc / calloc, NULL where the reference leaves NULL). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gmspipsio.h"
#include "pips_hip.h"

static double* take_vector(void* b, int which, int* len_out) {
   int len = 0;
   if (pips_gdx_block_vector(b, which, NULL, 0, &len)) return NULL;
   double* v = (double*)calloc(len > 0 ? (size_t)len : 1, sizeof(double));
   if (v && len > 0 && pips_gdx_block_vector(b, which, v, len, &len)) { free(v); return NULL; }
   if (len_out) *len_out = len;
   return v;
}

static int16_t* take_indicator(void* b, int which) {
   int len = 0;
   double* v = take_vector(b, which, &len);
   if (!v) return NULL;
   int16_t* out = (int16_t*)calloc(len > 0 ? (size_t)len : 1, sizeof(int16_t));
   for (int i = 0; out && i < len; ++i) out[i] = (int16_t)(v[i] != 0.0);
   free(v);
   return out;
}

static int take_matrix(void* b, int which, long long nnz, int32_t** rm, int32_t** ci, double** val) {
   int present = 0, rows = 0, cols = 0;
   *rm = NULL; *ci = NULL; *val = NULL;
   if (pips_gdx_block_matrix(b, which, &present, &rows, &cols, NULL, NULL, NULL)) return 1;
   if (!present) return 0;
   *rm = (int32_t*)calloc((size_t)rows + 1, sizeof(int32_t));
   *ci = (int32_t*)calloc(nnz > 0 ? (size_t)nnz : 1, sizeof(int32_t));
   *val = (double*)calloc(nnz > 0 ? (size_t)nnz : 1, sizeof(double));
   if (!*rm || !*ci || !*val) return 1;
   return pips_gdx_block_matrix(b, which, NULL, NULL, NULL, (int*)*rm, (int*)*ci, *val);
}

int readBlock(const int numBlocks, const int actBlock, const int strict, const int offset, const char* gdxFilename,
              const char* GAMSSysDir, GMSPIPSBlockData_t* blk) {
   (void)strict;       /* the reference's debug switch */
   (void)GAMSSysDir;   /* no GAMS installation needed */
   void* b = NULL;
   if (pips_gdx_read_block(&b, gdxFilename, numBlocks, actBlock, offset)) {
      printf("Could not read GDX file %s: %s\n", gdxFilename, pips_hip_last_error());
      return 1;
   }
   long long cnt[14];
   pips_gdx_block_counts(b, cnt);
   memset(blk, 0, sizeof(*blk));
   blk->numBlocks = numBlocks; blk->blockID = actBlock;
   blk->n0 = (int32_t)cnt[0]; blk->ni = (int32_t)cnt[1];
   blk->mA = (int32_t)cnt[2]; blk->mC = (int32_t)cnt[3]; blk->mBL = (int32_t)cnt[4]; blk->mDL = (int32_t)cnt[5];
   blk->nnzA = cnt[6]; blk->nnzB = cnt[7]; blk->nnzC = cnt[8]; blk->nnzD = cnt[9]; blk->nnzBL = cnt[10]; blk->nnzDL = cnt[11];
   int rc = 0;
   blk->c = take_vector(b, 0, NULL); blk->xlow = take_vector(b, 1, NULL); blk->xupp = take_vector(b, 2, NULL);
   blk->ixlow = take_indicator(b, 3); blk->ixupp = take_indicator(b, 4);
   blk->ixtyp = (int16_t*)calloc(blk->ni > 0 ? (size_t)blk->ni : 1, sizeof(int16_t));   /* all continuous */
   if (blk->mA) blk->b = take_vector(b, 5, NULL);
   if (blk->mC) { blk->clow = take_vector(b, 6, NULL); blk->cupp = take_vector(b, 7, NULL); blk->iclow = take_indicator(b, 8); blk->icupp = take_indicator(b, 9); }
   if (blk->mBL) blk->bL = take_vector(b, 10, NULL);
   if (blk->mDL) { blk->dlow = take_vector(b, 11, NULL); blk->dupp = take_vector(b, 12, NULL); blk->idlow = take_indicator(b, 13); blk->idupp = take_indicator(b, 14); }
   rc |= take_matrix(b, 0, cnt[6], &blk->rmA, &blk->ciA, &blk->valA);
   rc |= take_matrix(b, 1, cnt[7], &blk->rmB, &blk->ciB, &blk->valB);
   rc |= take_matrix(b, 2, cnt[8], &blk->rmC, &blk->ciC, &blk->valC);
   rc |= take_matrix(b, 3, cnt[9], &blk->rmD, &blk->ciD, &blk->valD);
   rc |= take_matrix(b, 4, cnt[10], &blk->rmBL, &blk->ciBL, &blk->valBL);
   rc |= take_matrix(b, 5, cnt[11], &blk->rmDL, &blk->ciDL, &blk->valDL);
   pips_gdx_block_destroy(b);
   return rc;
}
