"""MI355X-native path-tracing backend for the scenes and semantics of Rice-Rocket/rust-raytracer.

The directory name carries a hyphen, so it is imported through the `rust_raytracer_amd`
shim module at the repository root (or importlib); inside the package use relative imports.

  abi     ctypes mirror of include/rr_hip.h and include/rr_host.h
  host    scene construction (librr_host.so, CPU only)
  device  the HIP path behind the C ABI (librr_hip.so); fails loudly without it
  build   compiles both libraries in-tree
"""
from . import abi  # noqa: F401
