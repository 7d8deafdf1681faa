"""Drop-in for the reference's `simple_knn` package (simple-knn/setup.py); see `_C`."""
