"""Data preparation around the hot path (flowmap/misc/)."""
