"""MI355X-native hot path for 3DeeCellTracker-style tracking."""
__version__ = "0.1.0"
