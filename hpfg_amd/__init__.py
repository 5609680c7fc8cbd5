"""hpfg_amd: MI355X-native training hot path for HPFG-style semi-supervised segmentation."""
__version__ = "0.1.0"
