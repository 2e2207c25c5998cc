"""HIP-backed kernel provider with the reference's fortran_kernels.loader API."""
