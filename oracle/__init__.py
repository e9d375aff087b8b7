"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the GDMCF diffusion hot path (see oracle/gdmcf_oracle.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (gdmcf_amd) never does.
"""
