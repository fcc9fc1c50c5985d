// Internal interfaces between the translation units of the coordinate phase (not part of the C
// ABI): the stand-alone entry points and the fused frame front (front.hip) run the same kernels;
// the structs below carry the optional work a PRODUCER kernel does on behalf of a later stage, so
// that the frame front needs fewer launches (a launch costs 4-5 us on this part whatever it does).
#pragma once
#include "sf_common.h"

// columns [k0, k1) of a neighbour table := -1 for the rows in use (filled before the kernels that
// scatter into the table; a producer's launch can take the fill of a later table along)
struct SfNbrPrefill {
  int *nbr;
  int stride, k0, k1;
};

struct SfVoxFuse {
  int tables_ready;   // the slot table was filled by the caller
};

int sf_int_hard_voxelize(int count, const float *const *points, const int *n, int f,
                         const float voxel_size[3], const float coors_range[6], int max_points,
                         int max_voxels, float *voxels, int *coors, int coors_stride, int batch_idx0,
                         int *num_points_per_voxel, float *mean, int mean_features, int mean_stride,
                         int *row_count_dev, int rows_capacity, int *overflow_dev,
                         void *const *workspaces, size_t workspace_bytes, hipStream_t st,
                         const SfVoxFuse *fuse);
// the slot table inside a voxelizer workspace (what a caller with tables_ready fills with
// (EMPTY, INT_MAX) pairs): pointer and number of 32-bit words
void sf_int_vox_table(void *workspace, int n, int max_points, uint32_t **table, size_t *words);

struct SfStridedFuse {
  int tables_ready;        // out_keys (EMPTY) and `first` (INT_MAX) were filled by the caller
  int *first;              // cap ints of the caller (NULL: carved from the workspace)
  SfNbrPrefill pf_next;    // a later table on the OUTPUT set (the stage's SubM table): filled by
                           // the same launch that fills this rulebook's own table
  int defer_scatter;       // do not launch the table scatter: the caller runs it (fused launch)
};
struct SfScatterJob {      // what the deferred scatter needs
  const int *indices, *n_dev;
  int n_bound;
  int geo[16];             // the Geo struct of rulebook.hip, opaque here
  const int *slots, *vals;
  int *nbr;
  int nbr_stride, nbr_rows, small;
};
int sf_int_strided_rulebook(const int *indices, const int *n_dev, int n_bound, int batch,
                            const int out_shape[3], const int ksize[3], const int stride[3],
                            const int padding[3], uint32_t *out_keys, int *out_vals,
                            int out_capacity, int *outids, int outids_rows, int *n_out_dev,
                            int *overflow_dev, int *nbr, int nbr_stride, void *workspace,
                            size_t workspace_bytes, hipStream_t st, const SfStridedFuse *fuse,
                            SfScatterJob *job);
// SubM 3x3x3 table of a set (columns 14..26 pre-filled unless prefilled != 0), optionally in one
// launch with a deferred table scatter
int sf_int_subm27(const int *outids, const int *n_dev, int n_bound, int batch, const int in_shape[3],
                  const uint32_t *keys, const int *vals, int capacity, int *nbr, int nbr_stride,
                  int prefilled, const SfScatterJob *job, hipStream_t st);
int sf_int_scatter(const SfScatterJob *job, hipStream_t st);

// sf_row_order_by_mask with the histogram tables zeroed by the caller
int sf_int_row_order(const int *nbr, int nbr_stride, int K, int k_first, const int *n_dev, int n_bound,
                     int *row_order, void *workspace, size_t workspace_bytes, int hist_ready,
                     hipStream_t st);
void sf_int_row_order_hist(void *workspace, int n_bound, uint32_t **hist, size_t *words);

// sf_hash_build with the keys filled by the caller (keys_ready) and, optionally, the fill of a
// table on the same rows in the same launch
int sf_int_hash_build(const int *indices, const int *n_dev, int n_bound, const int shape[3], int batch,
                      uint32_t *keys, int *vals, int capacity, int *overflow_dev, int keys_ready,
                      SfNbrPrefill pf, hipStream_t st);
