"""empty stand-in (import only)"""
