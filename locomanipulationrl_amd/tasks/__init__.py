"""locomanipulationrl_amd/tasks (MI355X loco-manipulation step engine)."""
