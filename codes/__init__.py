"""`codes` package name kept so that `from codes.models import ...` / `python3 train.py --config codes/<x>_config.json`
of the reference keep working unchanged."""
