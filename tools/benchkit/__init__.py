"""bench.py's parts (workloads / host / stream / tilesplit_bench / report); bench.py at the repository root is the entry point"""
