"""Minimal stand-in for the braceexpand package (import only on the KMeans path)."""


def braceexpand(pattern):
    yield pattern
