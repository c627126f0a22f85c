"""import-only stand-in for python-fire (gen_golden.py calls the Cli methods directly)."""


def Fire(component):
    raise RuntimeError("fire stand-in: call the Cli methods directly")
