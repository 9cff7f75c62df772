"""Networks executed by the HIP GEMM kernels (sequential.Sequential, q_network.QNetwork)."""
from agents_amd.networks import layers, network, q_network, sequential  # noqa: F401
