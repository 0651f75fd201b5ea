"""Host logic of spsnet_amd.pointnet2_stack (autograd adapters, modules, the overflow-and-retry protocol, state_dict
names) on CPU: the extension's 14 entry points are patched with the oracle stand-in and the reference-generated fixtures
tests/golden/stackmod_*.npz are replayed.  No GPU, no compute through the HIP library."""
import torch

from oracle import ref_harness_stack as H
from tests import stack_replay as R

CPU = torch.device("cpu")


def test_stack_sa_and_fp_modules_match_reference_fixture():
    with H.patched_build_package():
        R.replay_sa_fp(CPU)


def test_vector_pool_modules_match_reference_fixture():
    with H.patched_build_package():
        R.replay_vector_pool(CPU)


def test_neighbor_voxel_sa_module_matches_reference_fixture():
    with H.patched_build_package():
        R.replay_voxel_sa(CPU)
