"""Ragged-batch ("stack") set-abstraction ops behind the names of pcdet/ops/pointnet2/pointnet2_stack (PV-RCNN /
Voxel-RCNN style callers; IA-SSD / SPSNet use the batch package).

  pointnet2_stack_cuda  the extension's surface: all 14 functions over libspsnet_sa.so (the drop-in boundary)
  _ops                  one plain function per op (allocation, validation, the overflow-and-retry protocol)
  pointnet2_utils / voxel_query_utils       autograd adapters under the reference's names
  pointnet2_modules / voxel_pool_modules    SA / FP / vector-pool / neighbour-voxel modules, written from the interface
"""
