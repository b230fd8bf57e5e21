"""locomanipulationrl_amd/envs (MI355X loco-manipulation step engine)."""
