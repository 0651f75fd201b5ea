"""Ragged-batch ("stack") set-abstraction ops: mirror of pcdet/ops/pointnet2/pointnet2_stack (PV-RCNN / Voxel-RCNN style
callers; IA-SSD / SPSNet use the batch package).  pointnet2_stack_cuda = the extension's surface (all 14 functions), pointnet2_utils /
voxel_query_utils = the autograd layer above it, pointnet2_modules = the SA / FP / vector-pool modules."""
