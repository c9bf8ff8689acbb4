"""``models.losses`` of the ``src/`` layout."""
