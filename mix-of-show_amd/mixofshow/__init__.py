"""mixofshow — MI355X-native (gfx950) implementation of the Mix-of-Show hot path.

Keeps the reference's import paths and entry points (EDLoRAPipeline, EDLoRATrainer,
RegionallyT2IAdapterPipeline, LoRALinearLayer, the attention processors, update_quasi_newton) and
routes their arithmetic through hand-written HIP kernels in libmos_hip.so (see include/mos_hip.h).
There is no CPU fallback: calling a kernel-backed op without the library or with CPU tensors raises.
"""
__version__ = '0.1.0'
