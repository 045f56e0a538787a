"""buffalo_amd: MI355X-native ALS / BPRMF / WARP training core behind buffalo's accelerator boundary."""
__version__ = "0.1.0"
