"""CPU oracle for the ARM-Net hot path — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from the product package (arm-net_amd/).  See armnet_oracle.c.
"""
