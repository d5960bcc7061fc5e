"""alphazero_amd -- MI355X-native self-play engine behind the t0m1ab/alphazero plugin surface.

The package needs libaz_amd.so (HIP, gfx950); importing the engine without it raises ImportError.
"""
__version__ = "0.1.0"
