"""Drop-in for the reference's ``simple_knn`` extension package (import path only; see ``simple_knn/_C.py``)."""
