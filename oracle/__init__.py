"""TEST INFRASTRUCTURE: CPU restatement of the reference's MIPS path (mips_oracle.py / mips_oracle.c) and the
deterministic data generators (synth.py).  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline() only -- never by the product package."""
