"""torchio_amd — MI355X-native (gfx950) engine for TorchIO's augmentation hot path."""
__version__ = "0.1.0"
