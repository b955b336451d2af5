"""Mirror of the reference's ``plnn`` import path -- only what the hot path needs."""
