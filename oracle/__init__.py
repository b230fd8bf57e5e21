"""oracle (MI355X loco-manipulation step engine)."""
