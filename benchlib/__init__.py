"""The parts of bench.py (the driver's entry stays `python bench.py` at the repo root): common (plumbing, Job), cpu (CPU legs), roofline
(counters, clock, copy ceiling), kernels (workloads), files (whole-file legs), multi (multi-GPU legs), metric (the runs)."""
