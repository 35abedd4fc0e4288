"""bench.py's parts as an importable package (round 6; round-5 verdict, weak 10 / next 5e): the workload (argument parser,
model and batch construction), the timing rules (clock settle, windows), the CPU baselines, the committed evidence under
profiles/ and the error line.  `bench.py` at the repository root is the driver's entry point: it holds `main()` — the
measurement flow of one rank — and re-exports these names, so `import bench` keeps working for the tests and tools."""
