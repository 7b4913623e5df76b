"""MI355X-native VoiceFixer 44.1 kHz inference hot path (host-side mirror of the reference surface)."""
__version__ = "0.1.0"
