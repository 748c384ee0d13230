"""MI355X-native alphaSurf hot path: sparse-voxel SDF ray-march + alpha compositing."""
__version__ = "0.1.0"
