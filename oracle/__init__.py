"""oracle/ -- CPU restatements of the reference step() path.  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product."""
