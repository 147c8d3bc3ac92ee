"""Drop-in mirror of the reference's `codes` package (codes/models.py, codes/trainers.py, codes/base.py,
codes/utils.py, codes/data_loader.py) on top of the HIP engine."""
