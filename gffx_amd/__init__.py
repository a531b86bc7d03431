"""gffx_amd -- MI355X-native engine for the `gffx intersect` hot path of Baohua-Chen/GFFx.

The package holds only what that path needs:

* ``csrc/``    hand-written HIP kernels for gfx950 + the C-ABI (``include/gffx_hip.h``) and the
               C++ host side that mirrors the reference's index_loader / commands::intersect;
* ``engine``   ctypes mirror of the C-ABI (device index, query batches, Join A / Join B);
* ``shard``    chromosome-bucket sharding of query batches across the GPUs of one node;
* ``synth``    seeded synthetic GFF3 / BED inputs (nothing real is available offline).

There is no CPU fallback: every compute entry point fails loudly when the HIP library or a GPU
is missing.
"""
__version__ = "0.1.0"
