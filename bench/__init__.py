"""bench.py's legs (repo root: bench.py is the entry point and holds only main())."""
