"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference algorithm (mmae_oracle.py, staging_oracle.py), the loader
for the read-only reference checkout (ref_loader.py) and the generator of the committed fixtures (make_golden.py).
Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package."""
