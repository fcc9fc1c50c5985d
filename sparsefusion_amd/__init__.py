"""sparsefusion_amd: MI355X-native LiDAR hot path of SparseFusion (voxelize -> SparseEncoder)."""
