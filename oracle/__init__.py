"""CPU oracle for the RepSurf-U hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under repsurf_amd/ does (tests/test_layout.py enforces it)."""
