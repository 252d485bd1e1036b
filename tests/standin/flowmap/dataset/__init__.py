"""(stand-in package: tests/standin/README.md)"""
