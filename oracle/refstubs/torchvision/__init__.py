"""Import-only stand-in for torchvision (absent here); see oracle/make_golden.py."""
