"""block2-preview_amd — MI355X-native H·psi path behind block2's plan-replay boundary.

Only what the hot path needs lives here: csrc/ (HIP kernels + the C ABI of include/b2x.h),
the ctypes binding (capi), the host-side mirror of the reference interface (batch_gemm, davidson,
parallel) and plan-file plumbing.  The HIP library is mandatory: nothing here computes on the CPU.
"""
__version__ = "0.1.0"
