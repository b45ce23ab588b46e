"""Stub: the reference's plots.py does `from IPython.display import clear_output`."""
