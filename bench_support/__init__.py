"""Legs of bench.py (the measurement tool at the repository root).  Not part of the product package."""
