// dense_chol.h -- hand-written dense SPD solve on the device (dense_chol.hip)
#pragma once
#include "corb_internal.h"
// Solves A x = b: A is n x n, row-major with leading dimension ld, symmetric positive definite; only its lower triangle is read, and its strictly-lower part is
// overwritten by the Cholesky factor's (A = L L'); the 32 x 32 diagonal blocks of A keep their input values -- their factors live in diag_ws
// (corb_chol_workspace_doubles(n) doubles of device memory, the caller's).  b[n]: right-hand side in, solution out.  *info (device): 0, or 1 + the first column with a
// non-positive pivot.  Everything is enqueued on `s`; nothing is read back.
size_t corb_chol_workspace_doubles(int n);
void corb_launch_chol_solve(double* A, int n, int ld, double* b, int* info, double* diag_ws, hipStream_t s);
