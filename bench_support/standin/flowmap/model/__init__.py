"""(stand-in package: bench_support/standin/README.md)"""
