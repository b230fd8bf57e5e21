"""locomanipulationrl_amd/train (MI355X loco-manipulation step engine)."""
