"""deepwmh_amd: MI355X-native nnU-Net 3d_fullres hot path for DeepWMH (HIP/gfx950 behind a C ABI)."""
__version__ = "0.1.0"
