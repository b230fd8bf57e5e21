"""locomanipulationrl_amd/utils (MI355X loco-manipulation step engine)."""
