"""upnerf_amd: MI355X-native (gfx950) implementation of the UP-NeRF render_rays training hot path.

Importing the sub-modules that touch the GPU (rendering, nerf, ...) loads libupnerf_hip.so and fails loudly when
it is missing; `upnerf_amd.synth` and `upnerf_amd.parallel` are importable without it (host-only helpers)."""
__version__ = "0.1.0"
