/* ORACLE — test infrastructure, not product code.
 * Plain C restatement of the two quadratic loops of the reference's string tools, used to check the native
 * bit-parallel kernels at sizes where the Python oracle is too slow:
 *   ref_levenshtein_u32   ocr_agent/tools.py:69-83  (single-row DP, unit costs)
 *   ref_lcs_align_u32     ocr_agent/tools.py:465-493 (full LCS table, backtrack prefers i-1 on ties)
 * Built by oracle/build_c.py into oracle/_build/libtext_ref.so. */
#include <stdint.h>
#include <stdlib.h>

int64_t ref_levenshtein_u32(const uint32_t* a, int64_t n, const uint32_t* b, int64_t m) {
  int64_t* row = (int64_t*)malloc((size_t)(m + 1) * sizeof(int64_t));
  for (int64_t j = 0; j <= m; ++j) row[j] = j;
  for (int64_t i = 1; i <= n; ++i) {
    int64_t diag = row[0];
    row[0] = i;
    for (int64_t j = 1; j <= m; ++j) {
      int64_t keep = row[j];
      int64_t best = row[j] + 1;
      if (row[j - 1] + 1 < best) best = row[j - 1] + 1;
      if (diag + (a[i - 1] != b[j - 1]) < best) best = diag + (a[i - 1] != b[j - 1]);
      row[j] = best;
      diag = keep;
    }
  }
  int64_t d = row[m];
  free(row);
  return d;
}

int ref_lcs_align_u32(const uint32_t* bb, int64_t n, const uint32_t* w, int64_t m, int32_t* out) {
  int32_t* t = (int32_t*)calloc((size_t)(n + 1) * (size_t)(m + 1), sizeof(int32_t));
  const int64_t W = m + 1;
  for (int64_t i = 1; i <= n; ++i)
    for (int64_t j = 1; j <= m; ++j) {
      if (bb[i - 1] == w[j - 1]) t[i * W + j] = t[(i - 1) * W + j - 1] + 1;
      else t[i * W + j] = t[(i - 1) * W + j] > t[i * W + j - 1] ? t[(i - 1) * W + j] : t[i * W + j - 1];
    }
  for (int64_t i = 0; i < n; ++i) out[i] = -1;
  int64_t i = n, j = m;
  while (i > 0 && j > 0) {
    if (bb[i - 1] == w[j - 1]) { out[i - 1] = (int32_t)(j - 1); --i; --j; }
    else if (t[(i - 1) * W + j] >= t[i * W + j - 1]) --i;
    else --j;
  }
  free(t);
  return 0;
}
