"""Entry point mirroring the reference's ``main.py`` (``make demo``): runs the headless two-view demo."""
from apps import sfm_headless

if __name__ == "__main__":
    sfm_headless.main()
