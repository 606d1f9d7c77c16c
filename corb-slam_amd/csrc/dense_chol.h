// dense_chol.h -- hand-written dense SPD solve on the device (dense_chol.hip)
#pragma once
#include "corb_internal.h"
// Solves A x = b in place: A is n x n, row-major with leading dimension ld, symmetric positive definite; only its lower triangle is read, and it is overwritten
// by the Cholesky factor L (A = L L').  b[n]: right-hand side in, solution out.  *info (device): 0, or 1 + the first column with a non-positive pivot.
// Everything is enqueued on `s`; nothing is read back.
void corb_launch_chol_solve(double* A, int n, int ld, double* b, int* info, hipStream_t s);
