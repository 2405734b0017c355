"""Serial stub of pathos (TEST INFRASTRUCTURE, this container only)."""
