"""freddy_amd -- MI355X-native PQ / IVFADC / kNN-join search behind the FREDDY UDF surface.

Importing the package does not touch the GPU; `freddy_amd.gpu.load()` dlopens
libfreddy_gpu.so and raises if it is missing (there is no CPU fallback in the product).
"""
from . import gpu  # noqa: F401
