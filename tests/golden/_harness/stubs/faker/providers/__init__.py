"""Harness stub (tests/golden only): stands in for the `faker` package, which the
reference imports for debug name strings. Not reference code."""
