"""adcraft_amd - MI355X-native vectorised BiddingSimulation step engine.

    from adcraft_amd import BiddingSimulation, bidding_sim_creator        # drop-in single env
    from adcraft_amd.vector_env import BiddingSimulationVectorEnv          # N envs, one engine call per step
    from adcraft_amd import rust                                           # drop-in for adcraft.rust

The step path runs only as HIP kernels (adcraft_amd/csrc) behind the C ABI in include/adcraft_engine.h;
there is no CPU fallback.
"""
from .gymnasium_kw_env import BiddingSimulation, bidding_sim_creator  # noqa: F401

__all__ = ["BiddingSimulation", "bidding_sim_creator"]
