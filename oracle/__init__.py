"""CPU ORACLE package — test infrastructure only.

Nothing under causal_eeg_vae_amd/ imports this package.  Allowed importers: tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg (as the checker / reported baseline).

  oracle.c_oracle   ctypes view of oracle/libceeg_oracle.so (ceeg_oracle.c): the bit-exact parts
                    (fp32 VQ with the specified fmaf-chain order, Philox streams, uniform negatives)
  oracle.ref_torch  plain PyTorch/NumPy CPU restatement of the module-level semantics
                    (VectorQuantizer, gcn_norm/GCNConv, VGAE, InnerProductDecoder, train step)
"""
