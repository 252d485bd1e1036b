"""Stand-in for the reference package's module layout (tests/standin/README.md).  Test infrastructure."""
