"""locomanipulationrl_amd/model (MI355X loco-manipulation step engine)."""
